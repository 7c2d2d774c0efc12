#!/usr/bin/env python
"""Placeholder for the reference's training entry point (flags at
/root/reference/README.md:54-63).  The training step (losses, backward kernels,
DDP gradient all-reduce) is SURVEY.md §8(f) next #1 and is not built yet: the
MI355X path currently covers the eval-mode forward only, and says so instead of
silently training on a fallback."""
import sys

if __name__ == "__main__":
    sys.exit("train_codet.py: the --com disco training step is not built on the MI355X path yet "
             "(SURVEY.md §8(f) next #1); use tools/det/test_codet.py for the forward path.")
