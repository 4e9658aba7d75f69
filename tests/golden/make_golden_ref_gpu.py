"""Writes tests/golden/ref_gpu_pin.npz: the outputs of the REFERENCE'S OWN device code (oracle/_ref/
libtf_ops_ref_fma.so = tf_sampling_g.cu / tf_grouping_g.cu compiled unmodified for gfx950, oracle/Makefile `ref_gpu`)
on the case catalogue of tests/ref_cases.py.  Needs a GPU:

    gpurun -- 'python tests/golden/make_golden_ref_gpu.py gpurun_out/ref_gpu_pin.npz'   (then copy to tests/golden/)

For every case: SHA-1 of every input (so the CPU test knows it regenerated the same inputs) and of every output;
outputs under 64 KiB are stored whole.  tests/test_ref_golden_cpu.py checks oracle/sa_oracle.c against this file in
the CPU suite."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_cases as RC   # noqa: E402
import ref_gpu           # noqa: E402


def main(out):
    dev = torch.device("cuda:0")
    ref = ref_gpu.RefOps("fma")
    store = {}
    cases = {}
    cases.update(RC.sa_cases())
    cases.update(RC.full_depth_cases())
    cases.update({k: v for k, v in RC.f4_cases().items() if not k.startswith(("three_", "k_interpolate"))})
    for name, case in sorted(cases.items()):
        outs = RC.run_torch(ref, case, dev)
        store[name + "/in_sha1"] = np.array([RC.sha1(a) for a in case[2]])
        store[name + "/out_sha1"] = np.array([RC.sha1(o) for o in outs])
        for i, o in enumerate(outs):
            if o.nbytes <= 65536:
                store["%s/out%d" % (name, i)] = o
        print(name, [o.shape for o in outs])
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_gpu_pin.npz"))
