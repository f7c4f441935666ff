import numpy as np, sys
sys.path.insert(0, '.')
from raytracer_amd import abi, device, host
from oracle import loader
from tests import common
text = common.all_primitives_scene()
fast = host.HostScene(text=text)
lin = host.HostScene(text=text, kd_tree=False)
ds = device.DeviceScene(fast.desc_ptr)
d = fast.desc
eye = np.array(d.eye[:]); look = np.array(d.look[:]); u = np.array(d.u[:]); v = np.array(d.v[:])
W, H = 160, 120
rays = []
for py in range(H):
    for px in range(W):
        x = px / W - 0.5; y = py / H - 0.5
        dr = look + x * u + y * v
        dr = dr / np.sqrt(dr @ dr)
        rays.append(np.concatenate([eye, dr]))
rays = np.array(rays)
def cmp(rays, tag):
    ref = loader.intersect_batch(lin.desc_ptr, rays)
    got = ds.intersect(rays, None, abi.RT_TRAVERSE_FAST)
    bad = [k for k in range(len(rays)) if got[k].object_id != ref[k].object_id or got[k].t != ref[k].t]
    print(tag, len(rays), "bad", len(bad))
    for k in bad[:8]:
        print("  ", k, list(rays[k]), got[k].object_id, ref[k].object_id, got[k].t, ref[k].t)
    return ref
for level in range(3):
    ref = cmp(rays, "level %d" % level)
    nxt = []
    for k in range(len(rays)):
        if ref[k].object_id == 0: continue
        p = rays[k, :3]; dd = rays[k, 3:]
        N = np.array(ref[k].N[:]); Q = p + ref[k].t * dd
        minusD = -1 * dd
        cosV = N * (minusD @ N)
        sinV = cosV + dd
        r = cosV + sinV
        r = r / np.sqrt(r @ r)
        nxt.append(np.concatenate([Q, r]))
    rays = np.array(nxt)
