"""MI355X drop-in modules for jdmulligan/bayesian-inference: ``emulation``, ``log_posterior``, ``mcmc``.

Put this directory BEFORE the reference's ``src`` on ``sys.path`` / ``PYTHONPATH``: the three modules
above then resolve here, every other module (``data_IO``, ``steer_analysis``, ``plot_*``, ``helpers``,
``common_base``, ``preprocess_input_data``) resolves to the untouched reference package through the
extended package path below.  See INTEGRATION.md.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
