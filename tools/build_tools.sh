#!/bin/bash
# Builds the C++ probes under tools/ (*.bin: git-ignored, they travel to the GPU box with gpurun).
set -e
cd "$(dirname "$0")/.."
LIB=monkey-moore_amd/lib
python3 -c "import sys; sys.path.insert(0,'.'); from __graft_entry__ import load_package; load_package().build.build_all()"
g++ -std=c++17 -O2 -Iinclude benchmarks/bench_search_mi355x.cpp -L$LIB -lmonkey-core -lmmoore_hip -Wl,-rpath,'$ORIGIN/../monkey-moore_amd/lib' -pthread -o tools/bench_search_mi355x.bin
g++ -std=c++17 -O2 -Iinclude benchmarks/bench_engine_file.cpp -L$LIB -lmonkey-core -lmmoore_hip -Wl,-rpath,'$ORIGIN/../monkey-moore_amd/lib' -pthread -o tools/bench_engine_file.bin
g++ -std=c++17 -O2 -Iinclude tools/scan_probe.cpp -L$LIB -lmmoore_hip -Wl,-rpath,'$ORIGIN/../monkey-moore_amd/lib' -pthread -o tools/scan_probe.bin
g++ -std=c++17 -O2 -Iinclude tools/mmoore_search.cpp -L$LIB -lmonkey-core -lmmoore_hip -Wl,-rpath,'$ORIGIN/../monkey-moore_amd/lib' -pthread -o tools/mmoore_search.bin
hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe.bin
ls -la tools/*.bin
