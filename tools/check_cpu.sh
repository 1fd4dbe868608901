#!/bin/bash
# rebuild the CPU-side artefacts (oracle, one-lane emulation of the device code) and run the CPU test suite from the repo root
set -e
cd "$(dirname "$0")/.."
make -s -C oracle
make -s -C tests/emu
python -m pytest tests -x -q -m "not gpu" "$@"
