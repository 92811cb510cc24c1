"""Drop-in for the reference's `simple_knn` package (/root/reference/dgmesh/submodules/simple-knn)."""
