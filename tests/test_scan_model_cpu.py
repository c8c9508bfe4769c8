"""tools/mfma_filter_model.py: the matrix-product first stage proposed for the sampled scan, modelled
on the CPU, emits exactly the oracle's run list (comparator.rs:157-250 through ora_diagonal_runs_all_pairs) -- the sum of
the head rows' distances is a necessary condition and every survivor is verified cell by cell."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_matrix_filter_model_emits_the_oracles_runs():
    for heads in ("4", "3"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_filter_model.py"), "4", "6", heads],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        d = json.loads(out.stdout.strip().splitlines()[-1])
        assert d["identical_run_lists"] and d["runs_model"] == d["runs_oracle"] > 0
        assert d["pass_whole_window"] <= d["pass_exact_head_rows"] <= d["pass_sum_filter"] < 0.05
