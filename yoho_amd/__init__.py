"""yoho_amd - MI355X-native implementation of YOHO's 60-rotation descriptor path and
YOHO-O / YOHO-C estimators behind the reference's plug-in API.

Sub-modules that need the GPU library (``hip``, ``network``, ``extractor``, ``matcher``,
``estimator``, ``knn_search``, ``yoho_extract``) load ``libyoho_hip.so`` on first use and raise
if it is missing - there is no CPU fallback.  ``tables``, ``weights`` and ``synth`` are pure
host helpers.
"""
__version__ = "0.1.0"
