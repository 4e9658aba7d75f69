import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = lambda n: importlib.import_module("3dssd_amd." + n)
cfgs, syn = pkg("configs"), pkg("synthetic")
dev = torch.device("cuda:0")
arch = cfgs.KITTI_3DSSD_ARCH
torch.cuda.reset_peak_memory_stats()
base = torch.cuda.memory_allocated()
pipe = pkg("pipeline").SAPipeline(arch, syn.random_backbone_params(arch), dev, batch=8, points=16384, coalesce=16,
                                  max_translate_range=cfgs.KITTI_MAX_TRANSLATE_RANGE)
x = torch.from_numpy(np.stack([syn.frame_of("default", f, 16384) for f in range(8)])).to(dev)
for _ in range(4 * 16):
    pipe.submit(x, sync_source=False)
pipe.drain()
free, total = torch.cuda.mem_get_info()
print("slots %d | torch allocated %.2f GB, reserved %.2f GB, peak allocated %.2f GB | device in use %.2f of %.1f GB"
      % (pipe.nslots, (torch.cuda.memory_allocated() - base) / 1e9, torch.cuda.memory_reserved() / 1e9,
         torch.cuda.max_memory_allocated() / 1e9, (total - free) / 1e9, total / 1e9))
