#!/bin/bash
cd "$(dirname "$0")" && "${HIPCC:-/opt/rocm/bin/hipcc}" --offload-arch=gfx950 -O2 -shared -fPIC spin.hip -o libspin.so && echo built libspin.so
