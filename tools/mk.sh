#!/bin/bash
# build libgsplat_hip.so from anywhere
make -C /root/repo/gaussiansplats3d_amd/csrc -j8 2>&1 | grep -E "error|warning: unused|Error" ; ls -la --time-style=+%T /root/repo/gaussiansplats3d_amd/csrc/libgsplat_hip.so | cut -c30-
