# Is the step-by-step fit of the small dense-gradient DeepFM reproducible run to run (float atomics in the dense table gradient)?
# Compares eager vs eager, graphed vs graphed and eager vs graphed tables after the test's three epochs.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import test_compiled_gpu as T
df, y = T._frame(64 * 23 + 17)
def run(spe):
    dm = T._model('DeepFM')
    T._fit(dm, df, y, spe)
    return {n: p.detach().clone() for n, p in dm.model.named_parameters()}
def diff(a, b):
    out = {}
    for n in a:
        e = (a[n] - b[n]).abs()
        out[n] = (float(e.max()), int((e > 5e-5).sum()))
    worst = max(out.items(), key=lambda kv: kv[1][0])
    return worst
for i in range(4):
    e1, e2, g1, g2 = run(1), run(1), run(10), run(10)
    print(i, 'eager-eager', diff(e1, e2), '| graph-graph', diff(g1, g2), '| eager-graph', diff(e1, g1), flush=True)
