"""Importable alias of the (hyphenated) package directory:  ``import hdrsky_amd as hs``."""
import importlib
import sys

_pkg = importlib.import_module(
    "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd")
sys.modules[__name__] = _pkg
