#!/usr/bin/env python3
"""Same command line as the reference's src/Quade.py: `Quade.py -c Conf.txt [-i -h]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quade_amd.quade import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
