"""CPU oracle for the `--com disco` hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: /root/reference holds no source (the code lives in the
un-vendored `coperception` submodule, /root/reference/.gitmodules:1-3), no
tests and no golden vectors, so this restatement follows SURVEY.md Appendix A
and cannot be checked against the reference itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under disconet_amd/ imports it.
"""
