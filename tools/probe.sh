#!/bin/bash
# tools/probe.sh — what this box has: GPU visibility, host cores, and
# whether REAL Ceres could be built here as a second CPU baseline (it hard-requires Eigen3 and abseil,
# CMakeLists.txt:132-155 of the reference; SuiteSparse is optional).  Read-only; prints key=value lines,
# `--brief` prints one line.  bench.py records the brief form in cpu_baseline.sample.
BRIEF=0; [ "$1" = "--brief" ] && BRIEF=1
have() { command -v "$1" >/dev/null 2>&1 && echo yes || echo no; }
first() { find / -xdev \( -path /proc -o -path /sys \) -prune -o "$@" -print 2>/dev/null | grep -v "/torch/\|/site-packages/\|/dist-packages/" | head -1; }
NPROC=$(nproc 2>/dev/null || echo "?")
KFD=no; [ -e /dev/kfd ] && KFD=yes
GPUS=$( (rocminfo 2>/dev/null || /opt/rocm/bin/rocminfo 2>/dev/null) | grep -c "Name:.*gfx950")
EIGEN=$(first -name "Core" -path "*Eigen/Core")
ABSL=$(first -name "flat_hash_map.h" -path "*absl/container*")
ABSL_LIB=$(first -name "libabsl_base*")
CHOLMOD=$(first -name "cholmod.h")
GTEST=$(first -name "gtest.h" -path "*gtest/gtest.h")
CAN_BUILD=no
[ -n "$EIGEN" ] && [ -n "$ABSL" ] && [ -n "$ABSL_LIB" ] && CAN_BUILD=yes
if [ $BRIEF = 1 ]; then
  echo "nproc=$NPROC gfx950_agents=$GPUS eigen3=$([ -n "$EIGEN" ] && echo yes || echo no) abseil=$([ -n "$ABSL" ] && echo yes || echo no) suitesparse=$([ -n "$CHOLMOD" ] && echo yes || echo no) real_ceres_buildable=$CAN_BUILD"
  exit 0
fi
echo "nproc=$NPROC"
echo "mem_total_kb=$(awk '/MemTotal/ {print $2}' /proc/meminfo 2>/dev/null)"
echo "cpu_model=$(awk -F: '/model name/ {print $2; exit}' /proc/cpuinfo 2>/dev/null | sed 's/^ //')"
echo "dev_kfd=$KFD"
echo "gfx950_agents=$GPUS"
echo "cmake=$(have cmake) ninja=$(have ninja) gcc=$(have gcc) hipcc=$(have hipcc)"
echo "eigen3_core=${EIGEN:-absent}"
echo "abseil_header=${ABSL:-absent}"
echo "abseil_library=${ABSL_LIB:-absent}"
echo "suitesparse_cholmod=${CHOLMOD:-absent}"
echo "gtest=${GTEST:-absent}"
echo "real_ceres_buildable=$CAN_BUILD"
[ $CAN_BUILD = no ] && echo "note=the reference cannot be built here: cpu_baseline.kind stays \"port\" (oracle/), no oracle/_ref"
exit 0
