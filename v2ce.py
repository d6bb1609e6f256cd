#!/usr/bin/env python3
"""Drop-in command line for the reference's ``v2ce.py`` (same flags, same npz output); the
implementation lives in ``v2ce-toolbox_amd/v2ce.py``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from v2ce_toolbox_amd.v2ce import main  # noqa: E402

if __name__ == "__main__":
    main()
