/* native_step.c - the C-ABI of include/wildfire.h used from plain C: no Python, no torch, no HIP headers.
 *
 * Steps `batch` 14x14 environments for `steps` ticks with pseudo-random N/S/E/W actions through the synchronous
 * host-buffer entry points (the shape of the reference's only FFI, pyastar/pyastar.py:7-22: caller-owned buffers,
 * integer status codes), prints a few aggregate numbers and a checksum of the final state.
 *
 *   gcc -O2 -Iinclude examples/native_step.c -o /tmp/native_step -Lwildfire_control_python_amd -lwildfire_hip \
 *       -Wl,-rpath,$PWD/wildfire_control_python_amd
 *   /tmp/native_step [batch] [steps]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wildfire.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != WF_OK) {                                                           \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, wf_strerror(rc_));       \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char **argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 100;
    const int H = 14, W = 14;
    wf_params p;
    wf_default_params(&p);
    p.auto_reset = 1;
    p.fuel_lo = 10;
    p.fuel_hi = 40;
    wf_ctx *ctx = NULL;
    CHECK(wf_create(&ctx, 0, batch, H, W, &p));
    printf("abi %d, kernel family: %s\n", wf_abi_version(), wf_kernel_name(ctx));
    CHECK(wf_reset(ctx, NULL, 1234));

    int8_t *actions = malloc((size_t)batch);
    double *reward = malloc(sizeof(double) * (size_t)batch);
    uint8_t *done = malloc((size_t)batch);
    int8_t *obs = malloc((size_t)batch * W * H * 3);
    int8_t *planes = malloc((size_t)batch * H * W * 4);
    wf_env *envs = malloc(sizeof(wf_env) * (size_t)batch);
    if (!actions || !reward || !done || !obs || !planes || !envs) return 3;

    uint32_t lcg = 12345u;
    double reward_sum = 0;
    long finished = 0;
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < batch; ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            actions[i] = (int8_t)(lcg >> 30);                /* 0..3 = N,S,E,W */
        }
        CHECK(wf_step_obs(ctx, actions, 1, 1ull, reward, done, obs));   /* [state, reward, done] of forest_fire.py:46-49 */
        for (int i = 0; i < batch; ++i) {
            reward_sum += reward[i];
            finished += done[i];
        }
    }
    CHECK(wf_download(ctx, planes, envs));
    uint64_t sum = 1469598103934665603ull;                   /* FNV-1a over the final planes */
    for (size_t i = 0; i < (size_t)batch * H * W * 4; ++i) sum = (sum ^ (uint8_t)planes[i]) * 1099511628211ull;
    long agents = 0, fire = 0;
    for (size_t i = 0; i < (size_t)batch * W * H; ++i) {
        agents += obs[3 * i];
        fire += obs[3 * i + 1];
    }
    printf("%d envs x %d steps: reward sum %.1f, episodes finished %ld, burning cells now %ld, visible agents %ld\n",
           batch, steps, reward_sum, finished, fire, agents);
    printf("state checksum %016llx\n", (unsigned long long)sum);
    CHECK(wf_destroy(ctx));
    free(actions); free(reward); free(done); free(obs); free(planes); free(envs);
    return 0;
}
