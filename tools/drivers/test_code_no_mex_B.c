/* The reference's no-mex driver (test_code_no_mex_B.c:383-387) on libldpc_hip: plain C,
 * built with gcc against include/ldpc_hip.h, no HIP headers.  The reference's main() calls
 *   test_code_B(30, 1, 0.01, "4095.737.3.txt", out)
 * and prints nothing (its printf lines are commented out, :377-380); this one takes the
 * same four values from the command line (defaults: the reference's) and prints out[].
 *   test_code_no_mex_B [H_file [iterations [trials [p_bsc]]]]                            */
#include <stdio.h>
#include <stdlib.h>

#include "ldpc_hip.h"

int main(int argc, char **argv) {
  const char *h_file = argc > 1 ? argv[1] : "4095.737.3.txt";
  const int iteration = argc > 2 ? atoi(argv[2]) : 30;
  const int trials = argc > 3 ? atoi(argv[3]) : 1;
  const double p = argc > 4 ? atof(argv[4]) : 0.01;
  double out[2];
  const int rc = test_code_B(iteration, trials, p, h_file, out);
  if (rc) {
    fprintf(stderr, "test_code_B: %s (%s)\n", ldpc_strerror(rc), ldpc_last_error());
    return 1;
  }
  printf("Errors = %f\nUndetected Errors = %f\n", out[0], out[1]);
  return 0;
}
