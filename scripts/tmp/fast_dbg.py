import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracer_amd import demo
scene = demo.DemoScene(0); ctx = scene.ctx
w, h = 240, 135
for frame in (0, 45, 95):
    out = {}
    for fast in (0, 1):
        ctx.set_option("fast", fast)
        rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda:0")
        hid = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
        ht = torch.empty((h, w), dtype=torch.float32, device="cuda:0")
        scene.render(frame, w, h, rgb, hit_id=hid, hit_t=ht); ctx.sync()
        out[fast] = (hid.cpu().numpy(), ht.cpu().numpy())
    d = out[0][0] != out[1][0]
    tie = d & (np.abs(out[0][1] - out[1][1]) <= 4e-6 * np.abs(out[0][1]))
    ys, xs = np.nonzero(d & ~tie)
    print(frame, int(d.sum()), int(tie.sum()), [(int(x), int(y), hex(int(out[0][0][y, x]) & 0xffffffff), hex(int(out[1][0][y, x]) & 0xffffffff), float(out[0][1][y, x]), float(out[1][1][y, x])) for x, y in zip(xs, ys)])
scene.close()
