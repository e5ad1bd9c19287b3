// How often does the double-double cube used on the device differ from glibc pow(x, 3.0)?  gcc -O2 -ffp-contract=off cube_vs_pow.c -lm
// (20M samples: 0.08 % differ, by 1 ulp; plain x*x*x differs in 26 %).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static double cube_dd(double x) {
  double p = x * x, e = fma(x, x, -p);
  double q = p * x, e2 = fma(p, x, -q);
  return q + (e2 + e * x);
}
int main() {
  long mism_dd = 0, mism_plain = 0, n = 20000000;
  srand48(1);
  for (long i = 0; i < n; i++) {
    double x = (i & 1) ? 6.3e6 + 1.0e6 * drand48() : exp(40 * (drand48() - 0.5));
    double r = pow(x, 3.0);
    if (cube_dd(x) != r) mism_dd++;
    if (x * x * x != r) mism_plain++;
  }
  printf("n=%ld  dd-cube != pow: %ld   x*x*x != pow: %ld\n", n, mism_dd, mism_plain);
  return 0;
}
