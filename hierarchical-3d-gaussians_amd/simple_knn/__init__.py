"""Drop-in ``simple_knn`` (reference: submodules/simple-knn, .gitmodules:7-9)."""
