import sys; sys.path.insert(0, '.')
import numpy as np, torch
from tests.util import SyntheticData
from tests.test_gpu_step import settings, make
from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
B, L, F = 64, 10, 136
data = SyntheticData(200, L, F, seed=3, pad_frac=0.1)
exp = settings('DLA', '', 'hidden_layer_sizes=[512, 256, 128],compute_dtype=bf16', Lmax=L, cutoff=L)
m, feed = make('DLA', data, exp, B)
fd, _ = feed.get_batch(data, check_validation=True)
m.step_local(fd)
torch.cuda.synchronize()
ws = m.model._ws[('fused', B, L)]
off = 87040 + 327680
wp = ws[off:off + 36864].view(torch.bfloat16).float().cpu().numpy().reshape(36, 512, 4)   # [k/4][n][k%4]
W0 = m.model.unflatten(m.model.params)[0]['W']
exp_ = np.zeros((144, 512), np.float32); exp_[:136] = W0
exp_ = exp_.reshape(36, 4, 512).transpose(0, 2, 1)
print('max abs diff of packed copy', np.abs(wp - exp_).max(), 'nonzero', (wp != 0).sum(), 'nan', np.isnan(wp).sum())
print(wp[0, :2], exp_[0, :2])
