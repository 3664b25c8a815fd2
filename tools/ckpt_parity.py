#!/usr/bin/env python
"""Launcher: checkpoint parity against the reference / oracle in one command -- the implementation is test infrastructure
(it runs the checker under oracle/) and lives in tests/ckpt_parity.py; see its docstring for the options.

    python tools/ckpt_parity.py --ckpt imagenet_k600.ckpt --images DIR [--frames 17] [--limit 16]
"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "ckpt_parity.py"), run_name="__main__")
