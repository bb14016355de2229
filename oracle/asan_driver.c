/* Sanitizer driver for the parity oracle (test infrastructure): a few hundred steps of
 * random actions with resets, built with -fsanitize=address,undefined by
 * tests/test_host_cpu.py::test_oracle_is_clean_under_asan_ubsan. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wf_oracle.h"

int main(void)
{
    const int shapes[4][2] = { { 10, 10 }, { 14, 14 }, { 23, 9 }, { 8, 40 } };
    unsigned rng = 12345u;
    for (int s = 0; s < 4; ++s) {
        const int H = shapes[s][0], W = shapes[s][1], B = 37;
        wfo_params p;
        wfo_default_params(&p);
        p.fuel_lo = 2; p.fuel_hi = 9; p.ignite_ppm = 30000; p.allow_dig_toggle = s & 1;
        int8_t *planes = calloc((size_t)B * H * W * 4, 1);
        wfo_env *envs = calloc((size_t)B, sizeof(wfo_env));
        int8_t *act = malloc((size_t)B), *obs = malloc((size_t)B * W * H * 3);
        double *rew = malloc(sizeof(double) * (size_t)B);
        uint8_t *done = malloc((size_t)B);
        if (wfo_reset(B, H, W, planes, envs, NULL, 7, 1000, &p)) return 2;
        for (int t = 0; t < 300; ++t) {
            for (int b = 0; b < B; ++b) { rng = rng * 1664525u + 1013904223u; act[b] = (int8_t)((rng >> 24) % 7) - 1; }
            if (wfo_step(B, H, W, planes, envs, act, t % 3 != 0, &p, rew, done)) return 3;
            if (wfo_get_state(B, H, W, planes, obs)) return 4;
            if (t % 50 == 49 && wfo_reset(B, H, W, planes, envs, done, 7 + t, 1000, &p)) return 5;
        }
        free(planes); free(envs); free(act); free(obs); free(rew); free(done);
    }
    int dx[32], dy[32];
    if (wfo_circle_points(3, dx, dy) != 16) return 6;
    puts("oracle clean");
    return 0;
}
