"""Where iterative_bubble_popping spends its time at a given size (row f1): cProfile of the second of two calls.
usage: bubbles_at_size.py N L V K [ERR]"""
import cProfile, contextlib, io, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from amira_amd import graph_utils as gu

N, L, V, K = (int(x) for x in sys.argv[1:5])
err = float(sys.argv[5]) if len(sys.argv) > 5 else 0.02
t = time.perf_counter()
calls, pos, fq = bench._bubble_inputs(4242, N, L, V, err)
print(f"inputs: {time.perf_counter() - t:.2f} s", flush=True)
n_windows = sum(max(0, len(v) - K + 1) for v in calls.values())


ARRAYS = os.environ.get("PROBE_ARRAYS") == "1"
if ARRAYS:
    reads_t, pos_t = gu._tokenized(calls, pos)


def drive():
    short, short_pos = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        if ARRAYS:
            a, b = reads_t, pos_t
        else:
            a, b = {r: list(v) for r, v in calls.items()}, {r: list(v) for r, v in pos.items()}
        t = time.perf_counter()
        reads, positions = gu.iterative_bubble_popping(a, b, 3, K, 1, short, short_pos, fq, tmp, 3, set(), 2)
        dt = time.perf_counter() - t
        n = int(reads.settled().read_offsets[-1]) if ARRAYS else sum(len(v) for v in reads.values())
        return dt, n


with contextlib.redirect_stderr(io.StringIO()):
    t0, _ = drive()
    pr = cProfile.Profile()
    pr.enable()
    t1, genes = drive()
    pr.disable()
print(f"arrays={ARRAYS} N={N} L={L} V={V} k={K}: first call {t0:.2f} s, second {t1:.2f} s under cProfile, {n_windows / t1 / 1e6:.3f} M gene-mers/s, genes out {genes}")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:6000])
