#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd databend_amd/host && timeout 60 ./host_selftest 2>&1 | tail -12
