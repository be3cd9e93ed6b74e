"""usage: python tools/api_profile.py [sort]  — cProfile of bench.py's api_e2e leg (graph_utils.cleaning_sweep's call
sequence through the Python drop-in on the cfg 3 stream; six sweeps: three warm + three timed), top entries."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

w = bench.WORKLOADS["cfg3-sweep"]
vocab, toks, offs = bench.make_tokens(w, 0, w["N"])
n_windows = int((offs[1:] - offs[:-1] - w["k"] + 1).clip(min=0).sum())
bench.run_api_e2e(w, vocab, toks, offs, w["k"], n_windows, steps=1)   # engines pooled and grown
pr = cProfile.Profile()
pr.enable()
res = bench.run_api_e2e(w, vocab, toks, offs, w["k"], n_windows, steps=3)
pr.disable()
print({k: res[k] for k in ("ms_per_step", "stages_ms_per_step")})
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(32)
print(out.getvalue())
if os.environ.get("AMG_PROFILE_CALLERS"):
    out = io.StringIO()
    st = pstats.Stats(pr, stream=out)
    for name in os.environ["AMG_PROFILE_CALLERS"].split(","):
        st.print_callers(name)
    print(out.getvalue())
