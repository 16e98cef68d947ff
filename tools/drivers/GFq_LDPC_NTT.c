/* The reference's GF(2^p) program (main(), GFq_LDPC_NTT.c:641-703) on libldpc_hip: plain C,
 * same command line, same "converged." / "failed." line per experiment (:691-694).
 *   GFq_LDPC_NTT [-iter num] <lap|bsc> NoiseLevel ParityMatrix                          */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldpc_hip.h"

int main(int argc, char **argv) {
  int iteration = 1000; /* :647 */
  int32_t verdict[3];
  int j, rc;
  if (argc == 6 && strcmp(argv[1], "-iter") == 0) {
    iteration = atoi(argv[2]);
    argc -= 2;
    argv += 2;
  }
  if (argc != 4) {
    fprintf(stderr, "usage: GFq_LDPC_NTT [-iter num] <lap|bsc> NoiseLevel ParityMatrix\n");
    return -1;
  }
  rc = GFq_LDPC_NTT(iteration, argv[1], atof(argv[2]), argv[3], verdict, NULL);
  if (rc) {
    fprintf(stderr, "GFq_LDPC_NTT: %s (%s)\n", ldpc_strerror(rc), ldpc_last_error());
    return -1;
  }
  for (j = 0; j < 3; j++) puts(verdict[j] ? "failed." : "converged.");
  return 0;
}
