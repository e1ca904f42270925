"""Drop-in alias of the reference package layout (run/train_sdm.py)."""
import sys
from lsdm_amd.run.train_sdm import main, build_parser, training_objective, validate  # noqa: F401

if __name__ == "__main__":
    sys.exit(main())
