import sys
from lsdm_amd.run.test_sdm import main, build_parser, HipBackend  # noqa: F401

if __name__ == "__main__":
    sys.exit(main())
