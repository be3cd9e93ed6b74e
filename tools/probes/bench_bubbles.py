"""bench.py's `bubbles` object alone (row f1)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print(json.dumps(bench.run_bubbles(0, with_cpu="--no-cpu" not in sys.argv), indent=1))
