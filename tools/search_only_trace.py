# phase times (NEEDLE_HIP_TRACE) of bench.py's search_only leg: warm calls of needle_audio_comparator_run(analyze=false)
# usage: python tools/search_only_trace.py [trace] [hostile] [episodes=280]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "trace" in sys.argv[1:]:
    os.environ["NEEDLE_HIP_TRACE"] = "1"
import bench
from needle_amd import capi, synth
hostile = "hostile" in sys.argv[1:]
episodes = next((int(a) for a in sys.argv[1:] if a.isdigit()), 280)
out = bench.search_only(capi, synth, episodes, 24.0, reps=6 if hostile else 20, hostile=hostile)
print("search_only", "hostile" if hostile else "tonal", "wall_ms", out["wall_ms"], "scan", out["scan_kernel_ms"], "form", out["scan_form"],
      "epilogue host fallbacks", out["epilogue_host_fallbacks"])
