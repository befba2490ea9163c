"""Dense X.W with K > 128 (the tiled MFMA kernel): tile heights.  SGX_XW_SHORT_TILES=1 selects the short tiles."""
import sys, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import ops
from tools.bench_configs import timed, rand_w
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
n = 232_965
X = torch.rand((n, 602), generator=gen, device="cuda").half()
W = rand_w(128, 602, gen)
print(json.dumps({"ms_xw_602_128": timed(lambda: ops.xw_dense(X, W), 50)}))
X32 = X.float(); W32 = W.float()
print(json.dumps({"ms_xw_602_128_fp32": timed(lambda: ops.xw_dense(X32, W32), 20)}))
W2 = rand_w(256, 602, gen)
print(json.dumps({"ms_xw_602_256": timed(lambda: ops.xw_dense(X, W2), 50)}))
G = torch.rand((2_449_029, 256), generator=gen, device="cuda").half()
W3 = rand_w(100, 256, gen)
print(json.dumps({"ms_xw_256_100_products_rows": timed(lambda: ops.xw_dense(G, W3), 20)}))
X4 = torch.rand((2_449_029, 100), generator=gen, device="cuda").half()
W4 = rand_w(256, 100, gen)
print(json.dumps({"ms_xw_100_256_products_rows": timed(lambda: ops.xw_dense(X4, W4), 20)}))
X5 = torch.rand((1 << 22, 64), generator=gen, device="cuda").half()
W5 = rand_w(64, 64, gen)
print(json.dumps({"ms_xw_64_64_s100m_rows": timed(lambda: ops.xw_dense(X5, W5), 20)}))
