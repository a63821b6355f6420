import sys; sys.path.insert(0,'.')
import torch
from tiny_diffusion_amd._lib import lib
torch.zeros(1,device='cuda')
for t in (64064,128064,128128): print(t, lib.tdx_diag_conv_occupancy(t))
p=torch.cuda.get_device_properties(0)
print(p.multi_processor_count, getattr(p,'max_threads_per_multi_processor',None), getattr(p,'shared_memory_per_multiprocessor',None), getattr(p,'shared_memory_per_block',None), getattr(p, 'shared_memory_per_block_optin', None))
