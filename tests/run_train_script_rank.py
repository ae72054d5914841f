"""Helper of tests/test_gpu_train_script.py: one RANK of the reference's train.py under the launcher when several
ranks have to share the test box's single GPU.  Test infrastructure, not part of the product: the product launch line is
`torchrun --nproc_per_node=N -m gangealing_amd.launch train.py ...` with one GPU per rank and RCCL.

Two things differ from that line, both because the box has ONE device: every rank uses cuda:0 (train.py:186-187 and
utils/distributed.py:6-14 take the device from LOCAL_RANK), and the process group is gloo (RCCL refuses two ranks on one
device; utils/distributed.py:11 hard-codes "nccl").  Everything else - rendezvous, the reference's own
DistributedDataParallel wrap of the STN and the latent learner (train.py:256-259), all_gather of the PCA latents,
reduce_loss_dict, rank-0-only checkpoints - is the script's own code."""
import os
import sys

import torch.distributed as dist

os.environ['LOCAL_RANK'] = '0'
_init = dist.init_process_group


def _init_gloo(backend=None, **kwargs):
    return _init(backend='gloo', **kwargs)


dist.init_process_group = _init_gloo

from gangealing_amd import launch          # noqa: E402

launch.main(sys.argv[1:])
