"""Drop-in for the reference's `Sim3DR` package (`from Sim3DR import RenderPipeline`, `get_normal`, `rasterize`): the same names,
served by the HIP kernels of synergynet_amd (no Cython extension to build)."""
from synergynet_amd.sim3dr import RenderPipeline, get_normal, rasterize  # noqa: F401
