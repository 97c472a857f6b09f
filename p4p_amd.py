"""Importable alias for the package whose directory name is not a Python identifier."""
import importlib
import sys

_pkg = importlib.import_module('adapting-2d-vits-for-3d-point-cloud-understanding_amd')
sys.modules[__name__] = _pkg
