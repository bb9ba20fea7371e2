import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, blu_amd, bench
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
cfg, B = sys.argv[1], int(sys.argv[2])
c = CONFIGS[cfg]; dev = torch.device("cuda", 0)
hs, ptrs, inputs, ms, nd, nnz = bench.batch_setup(c, B, dev, 0, blu_amd)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = blu_amd.factorize_batch(hs, device_ptrs=ptrs); torch.cuda.synchronize()
    print(os.environ.get("BLU_HIP_LIB", "new")[-12:], cfg, B, "rep", rep, "%.3f s" % (time.perf_counter() - t0), "launches", hs[0].stat(K.STAT_DEV_RELAUNCHES), "free %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9), flush=True)
