# phase times (NEEDLE_HIP_TRACE) of bench.py's search_only leg: warm calls of needle_audio_comparator_run(analyze=false)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "trace":
    os.environ["NEEDLE_HIP_TRACE"] = "1"
import bench
from needle_amd import capi, synth
print("search_only wall_ms", bench.search_only(capi, synth, 280, 24.0, reps=20)["wall_ms"])
