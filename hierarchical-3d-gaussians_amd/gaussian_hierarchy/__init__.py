"""Drop-in ``gaussian_hierarchy`` (reference: submodules/gaussianhierarchy, .gitmodules:10-12)."""
