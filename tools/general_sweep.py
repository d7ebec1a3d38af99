#!/usr/bin/env python
"""One pass over the common non-x2 factors (for a rocprofv3 --stats summary)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for dims in ("1280 720 1920 1080", "1280 720 3840 2160", "960 540 3840 2160", "2560 1440 3840 2160", "1920 1080 2560 1440", "3840 2160 1920 1080"):
    for pat in ("noise", "gradient"):
        env = dict(os.environ, NUS_PATTERN=pat)
        out = subprocess.run([sys.executable, os.path.join(here, "general_bench.py"), *dims.split(), "32"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if "us/frame" in line:
                print(pat, line)
