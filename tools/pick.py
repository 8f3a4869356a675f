#!/usr/bin/env python3
"""print chosen fields of bench.py's JSON line:  python tools/pick.py FILE key[.key] ..."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for path in sys.argv[2:]:
    v = d
    for k in path.split("."):
        v = v.get(k) if isinstance(v, dict) else None
    print(path, "=", v)
