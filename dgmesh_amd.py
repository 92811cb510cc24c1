"""Importable alias of the product package, whose directory name `dg-mesh_amd` is not a Python identifier."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("dg-mesh_amd")
