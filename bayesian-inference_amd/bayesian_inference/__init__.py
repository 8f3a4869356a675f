"""MI355X drop-in modules for jdmulligan/bayesian-inference: ``emulation``, ``log_posterior``, ``mcmc``.

Put this directory BEFORE the reference's ``src`` on ``sys.path`` / ``PYTHONPATH``: the three modules
above then resolve here, every other module (``data_IO``, ``steer_analysis``, ``plot_*``, ``helpers``,
``common_base``, ``preprocess_input_data``) resolves to the untouched reference package through the
extended package path below.  See INTEGRATION.md.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

# The reference's data_IO does ``from silx.io.dictdump import dicttoh5, h5todict`` (ref: data_IO.py:32); where silx
# is not installed those two names are served by gpemu.h5io (own HDF5 writer / reader, h5py if present), so that the
# untouched data_IO imports, reads observables.h5 and writes mcmc.h5.
from gpemu import h5io as _h5io  # noqa: E402

_h5io.install_silx_shim()
