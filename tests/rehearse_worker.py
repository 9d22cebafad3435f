"""One rank of tests/test_gpu_rccl.py::test_two_ranks_on_one_gpu_equal_single_process: started by
torch.distributed.run, every rank on cuda:0, gloo collectives (RCCL refuses two ranks on one device; the
collective then runs on a host copy, speaker-recognition_amd/dist.py).  Rank 0 writes what the drop-in classes
returned to the .npz named on the command line."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as td
    torch.cuda.set_device(0)
    td.init_process_group("gloo")
    import sidekit_amd as sk
    from test_gpu_rccl import _run_all
    assert sk.package.dist.is_dist() and sk.package.dist.world_size() == int(os.environ["WORLD_SIZE"])
    out = _run_all(sk)
    payload = dict(sk.package.dist.PAYLOAD)
    torch.cuda.synchronize()
    td.barrier()
    if td.get_rank() == 0:
        np.savez(sys.argv[1], payload_all_reduce=payload["all_reduce"], payload_all_gather=payload["all_gather"],
                 payload_reduce_scatter=payload["reduce_scatter"], **out)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
