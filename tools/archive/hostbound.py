import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from rvpt_amd import RVPT, scene, native
tris, mats = scene.default_scene()
for W,H in [(64,64),(256,256)]:
    r = RVPT(W, H, flags=native.TIMING)
    r.add_triangles(tris); [r.add_material(m) for m in mats]
    r.initialize()
    for _ in range(50): r.update(); r.draw()
    r.wait(); r.context.reset_timing()
    t0=time.perf_counter()
    for _ in range(1000): r.update(); r.draw()
    t1=time.perf_counter(); r.wait(); t2=time.perf_counter()
    _, ks, n = r.context.timing()
    print(W,H,"host issue us/frame", (t1-t0)*1e3, "total us/frame", (t2-t0)*1e3, "kernel us", ks/n*1e3)
    # only C calls
    ctx=r.context; s=r.render_settings.pack(); cam=r.scene_camera.get_data()
    t0=time.perf_counter()
    for i in range(1000): ctx.set_frame(s, cam); ctx.dispatch()
    t1=time.perf_counter(); r.wait(); t2=time.perf_counter()
    print("   ctypes-only: issue us/frame", (t1-t0)*1e3, "total", (t2-t0)*1e3)
    r.shutdown()
