"""The streaming decode kernel hides its loads from hipcc (inline asm) and releases them with counted waits;
nothing may touch a destination register in between (tools/audit_asm_loads.py explains the failure this guards
against).  Compiles the kernel instantiation units to assembly (gfx950 cross-compile, no GPU) and lints them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_an_in_flight_hidden_load():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("0 finding(s)") == 6, r.stdout[-2000:]
