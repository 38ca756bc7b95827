"""Developer probe: a few training steps of the guiding network on 524288 points (random or
pixel-ordered positions), meant to be run under rocprofv3 --kernel-trace --stats."""
import sys

import numpy as np

sys.path.insert(0, ".")
from elaina_amd.guided import GuidingNetwork  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "random"
n = 524288
rng = np.random.default_rng(0)
if mode == "random":
    xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
else:
    i = np.arange(n)
    xy = np.stack([(i % 1024 + 0.5) / 1024 * 0.7 + 0.15, (i // 1024 + 0.5) / 1024 * 0.7 + 0.15], 1).astype(np.float32)
dl = rng.normal(size=(n, 33)).astype(np.float32)
net = GuidingNetwork(seed=1)
for _ in range(4):
    net.train_step(xy, dl, apply_update=False)
net.close()
