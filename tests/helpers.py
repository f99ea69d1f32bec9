"""Test-side name of the checker harness (oracle/harness.py: the CPU oracle driven over engine state -- test infrastructure;
bench.py's `parity` / `cpu_baseline` legs and __graft_entry__.smoke() import it from there, not from the tests package)."""
from oracle.harness import *  # noqa: F401,F403
from oracle.harness import O  # noqa: F401
