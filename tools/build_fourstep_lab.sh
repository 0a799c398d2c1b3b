#!/bin/bash
# lab build of the library sources with phase stamps + tools/fourstep_lab.hip
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-value -DARMFFT_LAB_STAMPS -Iinclude -Ifft_amd/csrc -fgpu-rdc \
  -x hip tools/fourstep_lab.hip fft_amd/csrc/armfft_abi.cpp fft_amd/csrc/armfft_f32.hip fft_amd/csrc/armfft_fixed.hip fft_amd/csrc/armfft_large.hip \
  fft_amd/csrc/armfft_gnss.hip fft_amd/csrc/armfft_resamp.hip -o tools/fourstep_lab -lpthread
