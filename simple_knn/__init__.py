"""Drop-in module name for the reference's `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20)."""
