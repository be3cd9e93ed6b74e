import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for abl in ("0", "1", "2"):
    env = dict(os.environ, AMG_GAP_ABLATE=abl)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("AMG_GAP_ABLATE", abl, "correct_gapped ms/step", d["stages_ms_per_step"]["correct_gapped"])
