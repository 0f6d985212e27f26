#!/bin/bash
mkdir -p gpurun_out/r5h
timeout 600 python tools/probe/r5_e2e_profile.py B > gpurun_out/r5h/e2e_profile_B.log 2>&1
head -80 gpurun_out/r5h/e2e_profile_B.log | cut -c1-200
