/*
 * Build-time generator of theta_table.inc: theta(s) for every integer similarity s in [0, 512^2]
 * as float32 bit patterns, so that the pair kernel's angle / ratio tests are two table reads instead
 * of two double-precision acos evaluations per row and column (csrc/matcher.hip, accept_tab).
 *
 *   theta(s) = RN_f32(acos_f64(min(f32(s) * 2^-18, 1)))     (specification: oracle/matcher_oracle.py)
 *
 * The same expression runs on the device in theta_dev(); tests/test_matcher_gpu.py compares the
 * table (vc_theta_table), the device evaluation (vc_theta_eval) and the oracle for every input.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
  for (int s = 0; s <= 512 * 512; ++s) {
    float x = (float)s * (1.0f / (512.0f * 512.0f));
    if (x > 1.0f) x = 1.0f;
    const float t = (float)acos((double)x);
    uint32_t u;
    memcpy(&u, &t, 4);
    printf("0x%08xu,%s", u, (s % 8 == 7) ? "\n" : "");
  }
  printf("\n");
  return 0;
}
