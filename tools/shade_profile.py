"""Runs N frames of a workload and lets the context's destructor print k_shade's per-phase cycle profile
(a library built with -DRND_SHADE_PROFILE: tools/build_variant.sh prof -DRND_SHADE_PROFILE; RND_LIBRARY=build_ab/lib_prof.so)."""
import sys
sys.path.insert(0, "/root/repo")
import bench
from renderer_amd import VisualServer, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "c3_grid_10m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
scene = getattr(scenes, name)()
vs = VisualServer(scene.width, scene.height, device=0)
drv = bench.FrameDriver(vs, scene)
for _ in range(n):
    drv.frame()
vs.finish()
vs.close()
