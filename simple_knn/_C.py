"""`from simple_knn._C import distCUDA2` (KNN/ext.cpp:15-16) -> MI355X implementation in dg-mesh_amd/knn.py."""
import importlib

distCUDA2 = importlib.import_module("dg-mesh_amd.knn").distCUDA2
