#!/usr/bin/env python3
"""Differential fuzz of the pickle walker (`_cols_native.pickle_table`) against `pickle.loads`, beyond the test suite's 4 500 streams:
    python scripts/fuzz_walker.py [seed] [streams per protocol]
one to three random byte corruptions of a 300-row task list (x.5 positions, LONG1 lengths, memo-referenced and non-ASCII names) in
protocols 2-5, random size hints.  The walker may decline a stream (None / ValueError: `resolve._store_for` then lets pickle read the
block); one it accepts must be accepted by pickle too, and then give the SAME table.  No GPU involved.  Exits non-zero otherwise."""
import os
import pickle
import random
import signal
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutesv_amd import _cols_native as cn      # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = random.Random(seed)
rows = [(1000 + i * 7 + (0.5 if i % 9 == 0 else 0), 40 + i % 13 if i % 17 else (1 << 40) + i, ("read_%d_é" % i) if i % 50 == 0 else "read_%d" % (i // 2),
         "ACGT" * (i % 9 + 1), "INS", "1") for i in range(300)]


def on_alarm(*_):
    raise TimeoutError()


signal.signal(signal.SIGALRM, on_alarm)
both = walker_only = declined = differ = 0
for proto in (2, 3, 4, 5):
    blob = pickle.dumps(rows, protocol=proto)
    for _ in range(n_streams):
        b = bytearray(blob)
        for _k in range(rng.choice([1, 1, 1, 2, 3])):
            b[rng.randrange(len(b))] = rng.randrange(256)
        b = bytes(b)
        try:
            t = cn.pickle_table(b, 0, 6, (0, 1), (2, 3, 4, 5), rng.choice([-1, len(b), len(b) // 2, 7]))
        except ValueError:
            t = None
        if t is None:
            declined += 1
            continue
        signal.alarm(5)
        try:
            want = pickle.loads(b)
            ok = isinstance(want, list)
        except BaseException:                              # noqa: BLE001  (whatever pickle raises on garbage, a time-out included)
            ok = False
        finally:
            signal.alarm(0)
        if not ok:
            walker_only += 1
            continue
        both += 1
        same = t[0] == len(want) and np.frombuffer(t[2][0], np.int64).tolist() == [int(r[0]) for r in want] and \
            np.frombuffer(t[2][1], np.int64).tolist() == [int(r[1]) for r in want]
        for k, (o, ln) in zip((2, 3, 4, 5), t[3]):
            o, ln = np.frombuffer(o, np.int64), np.frombuffer(ln, np.int32)
            same = same and [b[int(x):int(x) + int(y)].decode("utf-8", "surrogatepass") for x, y in zip(o, ln)] == [r[k] for r in want]
        differ += not same
print("%d streams: %d accepted by both (%d with different tables), %d by the walker only, %d declined" % (4 * n_streams, both, differ, walker_only, declined))
sys.exit(1 if (walker_only or differ) else 0)
