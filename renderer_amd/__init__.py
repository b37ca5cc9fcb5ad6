"""renderer_amd — MI355X-native frame hot path behind the reference's VisualServer surface.

Only what the path needs: the C-ABI HIP library (csrc/), the host-side mirror of the reference
interface (visual_server.py, types.py) and the synthetic scenes of BASELINE.json (scenes.py).
"""
from ._capi import RND_NONE, RndError, load_product, product_library_path  # noqa: F401
from .types import (AssetServer, BillboardMode, Camera, Color, Handle, Light, Material, Mesh,  # noqa: F401
                    Submesh, VERTEX_DTYPE)
from .visual_server import ToneMapping, VisualServer  # noqa: F401
