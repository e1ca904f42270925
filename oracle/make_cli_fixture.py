"""Records tests/golden/g10_cli_ddim50.npz: BASELINE configs[0] -- `python -m run.test_sdm <root> --synthetic 4 --use_ddim
--timestep_respacing ddim50 --fix_ori --test_on_valid_set` driven by the CPU ORACLE (tests/test_host_logic.py::OracleBackend;
oracle/cpu_ref.py is pinned to goldens recorded from the real reference).  200 oracle forwards (about 10 minutes on 8 cores), which
is why the GPU suite compares the HIP CLI run with this recording instead of re-running the oracle on the GPU box.

    python oracle/make_cli_fixture.py            # g10_cli_ddim50 (B = 1, the reference's loop)
    python oracle/make_cli_fixture.py --b4       # g10b_cli_ddim5_b4: the same CLI with --batch_size 4 --timestep_respacing ddim5

Test infrastructure: nothing under lsdm_amd/ imports this.  The synthetic data set is regenerated from its seed by the CLI, the file
holds the expected outputs only (predictions, guiding points, results.txt)."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ARGV = ["--synthetic", "4", "--use_ddim", "--timestep_respacing", "ddim50", "--fix_ori", "--test_on_valid_set", "--model_name", "sdm"]


ARGV_B4 = ["--synthetic", "4", "--use_ddim", "--timestep_respacing", "ddim5", "--batch_size", "4", "--fix_ori", "--test_on_valid_set",
           "--model_name", "sdm"]


def main():
    b4 = "--b4" in sys.argv
    argv, name = (ARGV_B4, "g10b_cli_ddim5_b4.npz") if b4 else (ARGV, "g10_cli_ddim50.npz")
    import run.test_sdm as cli
    from test_host_logic import OracleBackend
    torch.set_num_threads(min(8, os.cpu_count()))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "out")
        rc = cli.main([os.path.join(tmp, "data"), "--device", "cpu", "--output_dir", out] + argv, backend_factory=OracleBackend)
        assert rc == 0
        rec = {"results_txt": np.array(open(os.path.join(out, "results.txt")).read())}
        for sub in ("predictions", "guiding_points"):
            files = sorted(os.listdir(os.path.join(out, sub)))
            rec[sub + "_files"] = np.array(files)
            rec[sub] = np.stack([np.load(os.path.join(out, sub, f)) for f in files])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), **rec)
    print(rec["results_txt"])


if __name__ == "__main__":
    main()
