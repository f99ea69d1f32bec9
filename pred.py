#!/usr/bin/env python
"""`python pred.py`: launcher of wide_deep_amd.cli.pred_main (flags, schedules and output format are documented there)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wide_deep_amd.cli import pred_main  # noqa: E402

if __name__ == "__main__":
    pred_main()
