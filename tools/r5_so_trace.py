# phase times (NEEDLE_HIP_TRACE) of bench.py's search_only leg: warm calls of needle_audio_comparator_run(analyze=false)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NEEDLE_HIP_TRACE"] = "1"
import bench
from needle_amd import capi, synth
print(bench.search_only(capi, synth, 280, 24.0, reps=3)["wall_ms"])
