import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
from _util import product_unet, tiles
B = int(sys.argv[1])
m = product_unet("uncond", 32, (1, 2, 4))
x = tiles(3, 6, 40).cuda()[:B].contiguous(); t = torch.full((B,), 17, device="cuda")
m(x, t); torch.cuda.synchronize()
