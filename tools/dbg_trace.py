"""Debug aid: edit scripts of a few simple pairs from both trace kernels next to the oracle's."""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from oraclelib import Oracle  # noqa: E402
from pacbioassembly_amd import engine as eng  # noqa: E402
from pacbioassembly_amd.engine import PAIR_DTYPE  # noqa: E402

ctx = eng.Context(0)
orc = Oracle()
rng = np.random.RandomState(1)
alpha = np.frombuffer(b"ACGT", np.uint8)


def mutate(a, e):
    out = []
    for ch in a:
        x = rng.rand()
        if x < e / 3:
            out += [alpha[rng.randint(4)], ch]
        elif x < 2 * e / 3:
            pass
        elif x < e:
            out.append(alpha[rng.randint(4)])
        else:
            out.append(ch)
    return np.array(out, np.uint8)


cases = []
for ln, e, tail in [(100, 0.0, 0), (100, 0.1, 30), (700, 0.15, 200), (700, 0.24, 200), (3000, 0.15, 500), (3000, 0.24, 500)]:
    a = alpha[rng.randint(0, 4, ln)]
    b = np.concatenate([mutate(a, e), alpha[rng.randint(0, 4, tail)]])
    cases.append((a.tobytes(), b.tobytes(), f"len{ln}_e{e}"))
    cases.append((b.tobytes(), a.tobytes(), f"len{ln}_e{e}_swapped"))

seqs, pairs = [], []
for a, b, _ in cases:
    pairs.append((len(seqs), 0, len(a), len(seqs) + 1, 0, len(b), 0))
    seqs += [a, b]
S = ctx.seqs_from_list(seqs, strict_acgt=True)
arr = np.array(pairs, PAIR_DTYPE)
for kernel in (1, 2):
    out, scripts = ctx.align_batch_trace(S, S, arr, 0.30, kernel=kernel)
    for (a, b, tag), got, ops in zip(cases, out, scripts):
        exp = orc.align(a, b, 0.30, want_ops=True)
        same = ops.tolist() == exp["ops"].tolist()
        print(f"kernel {kernel} {tag}: rc {got['rc']}/{exp['rc']} cost {got['cost']}/{exp['cost']} md {got['max_dst']} "
              f"nedit {ops.size}/{exp['nedit']} same={same}")
        if not same:
            e = exp["ops"]
            k = next((i for i in range(min(len(e), ops.size)) if e[i] != ops[i]), min(len(e), ops.size))
            print("   first difference at", k, "got", ops[max(0, k - 5):k + 25].tolist(), "exp", e[max(0, k - 5):k + 25].tolist())
            print("   got counts", np.bincount(ops, minlength=4).tolist(), "exp counts", np.bincount(e, minlength=4).tolist())
