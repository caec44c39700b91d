"""round 6: where do the small torch kernels (fills, copies, adds) of the layer-by-layer graphs come from?
Runs a few EAGER train steps of bench.py's AutoInt / xDeepFM configuration under torch.profiler with Python stacks and
prints, per small device kernel, the deeptables_amd frames that launched it.
usage: python tools/r6/glue_trace.py AutoInt|xDeepFM [steps]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else 'AutoInt'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from deeptables_amd.models import deepnets
    nets = {'xDeepFM': deepnets.xDeepFM, 'AutoInt': deepnets.AutoInt, 'DeepFM': deepnets.DeepFM}[model]
    device = torch.device('cuda:0')
    dim = 32 if model == 'AutoInt' else bench.D
    dm = bench.build_model(nets, device, None, dim, bench.MODEL_PARAMS.get(model))
    bench.N_BATCHES = 4
    batches = bench.make_batches(8192, device, seed=1)

    for i in range(3):
        bench_step(dm, batches[i % 4])
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for i in range(steps):
            bench_step(dm, batches[i % 4])
        torch.cuda.synchronize()
    ev = prof.events()
    # device kernels per launching CPU op: walk the CPU ops that have a stack, count their kernels
    agg = collections.Counter()
    dur = collections.Counter()
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
            continue
        for k in e.kernels:
            name = k.name
            if name.startswith('dt::') or 'dt::k_' in name:
                continue
            frames = [f for f in (e.stack or []) if 'deeptables_amd' in f or 'bench.py' in f][:4]
            key = (name[:70], e.name, tuple(str(s) for s in (e.input_shapes or [])[:3]), tuple(frames))
            agg[key] += 1
            dur[key] += k.duration
    print(f'== {model}: non-library device kernels of {steps} eager steps, by launching op and deeptables_amd frames')
    for key, n in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
        name, op, shapes, frames = key
        print(f'{n / steps:6.2f}/step {dur[key] / steps:8.1f} us/step  {name}\n        op {op} {shapes}')
        for f in frames:
            print('          ', f)



def bench_step(dm, b):
    idx, dense, y = b
    dm.train_step((idx, dense), y)


if __name__ == '__main__':
    main()
