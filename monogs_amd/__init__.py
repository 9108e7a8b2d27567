"""monogs_amd -- MI355X-native differentiable Gaussian rasteriser with camera-pose Jacobians,
a drop-in for MonoGS's ``diff_gaussian_rasterization`` and ``simple_knn``."""
__version__ = "0.1.0"
