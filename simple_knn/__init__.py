"""Drop-in package name for MonoGS: ``from simple_knn._C import distCUDA2``
(/root/reference/gaussian_splatting/scene/gaussian_model.py:18)."""
