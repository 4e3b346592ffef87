#!/bin/bash
# Builds scratch/proto/libgusto_oracle_seg.so: a GENERATED copy of the oracle whose riccati_factor / riccati_solve are renamed
# *_seq, with tools/proto/segriccati.c (the segmented KKT solve) appended.  Prototype only; the committed oracle is untouched.
set -e
cd "$(dirname "$0")/../.."
mkdir -p scratch/proto
sed -e 's/^static int riccati_factor(go_problem\* p, int ng, const int\* gidx) {$/static int riccati_factor(go_problem* p, int ng, const int* gidx);\nstatic int __attribute__((unused)) riccati_factor_seq(go_problem* p, int ng, const int* gidx) {/' \
    -e 's/^static void riccati_solve(go_problem\* p, int ng, const int\* gidx, const double\* rg) {$/static void riccati_solve(go_problem* p, int ng, const int* gidx, const double* rg);\nstatic void __attribute__((unused)) riccati_solve_seq(go_problem* p, int ng, const int* gidx, const double* rg) {/' \
    oracle/gusto_oracle.c > scratch/proto/gusto_oracle_seg.c
cat tools/proto/segriccati.c >> scratch/proto/gusto_oracle_seg.c
gcc -O2 -std=gnu99 -fPIC -Wall -Wextra -fopenmp -Ioracle -shared -o scratch/proto/libgusto_oracle_seg.so scratch/proto/gusto_oracle_seg.c -lm
echo built scratch/proto/libgusto_oracle_seg.so
