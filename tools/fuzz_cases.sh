#!/bin/bash
# Runs the listed cases of the randomised parity sweep through the full test logic (strict check, then the second look).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for s in "$@"; do python3 "$ROOT/tools/fuzz_case.py" "$s" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-400; done
