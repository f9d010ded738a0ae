/* Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizers.py; built against
 * lib/libuavac_asan.so = `make -C uav-autonomous-control_amd asan`).  Runs WITHOUT a GPU: everything an entry point does before
 * it needs a device -- null contexts, argument checks, the vehicle defaults, version queries, context creation failing
 * cleanly.  Exit code 0 = every call answered as documented; the sanitizers abort the process on their own findings. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "uavac.h"

#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { fprintf(stderr, "line %d: %s\n", __LINE__, #cond); return 1; } \
    } while (0)

int main(void) {
    EXPECT(uavac_version() == UAVAC_VERSION);
    uavac_vehicle V;
    memset(&V, 0xff, sizeof V);
    uavac_vehicle_default(&V);
    EXPECT(V.g == 9.81 && V.inner_per_outer == 10 && V.ground == 0 && V.ground_timeconst == 0.02);
    uavac_vehicle_default(NULL);
    int built = 0, rt = 0;
    EXPECT(uavac_comm_versions(&built, &rt) == UAVAC_OK && built >= 21000 && rt >= 21000);
    EXPECT(uavac_comm_versions(NULL, NULL) == UAVAC_OK);

    uavac_ctx *ctx = (uavac_ctx *)0x1;
    const int rc = uavac_create(&ctx, -1);
    if (rc == UAVAC_OK) {               /* a GPU is present after all: the GPU tests cover the rest */
        uavac_destroy(ctx);
        printf("asan driver: GPU present, context created and destroyed\n");
        return 0;
    }
    EXPECT(rc == UAVAC_EHIP && ctx == NULL);            /* no device: fails cleanly, never a CPU fallback */
    EXPECT(uavac_create(NULL, 0) == UAVAC_EINVAL);
    uavac_destroy(NULL);
    EXPECT(strcmp(uavac_last_error(NULL), "null context") == 0);
    EXPECT(strcmp(uavac_last_rollout_kernel(NULL), "") == 0);
    EXPECT(uavac_device(NULL) == UAVAC_EINVAL);
    EXPECT(uavac_set_stream(NULL, NULL) == UAVAC_EINVAL && uavac_reset_stream(NULL) == UAVAC_EINVAL);
    EXPECT(uavac_synchronize(NULL) == UAVAC_EINVAL && uavac_set_option(NULL, "log_pitch", 0) == UAVAC_EINVAL);
    int32_t flags[4];
    EXPECT(uavac_take_flags(NULL, flags) == UAVAC_EINVAL);

    /* every data entry point refuses a null context before it touches an argument */
    double d[64] = {0};
    int32_t i32[64] = {0};
    int64_t i64[8] = {0};
    EXPECT(uavac_minsnap_row_counts_dev(NULL, d, 1, 1, 1.0, 0.01, d, i32, i64) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_solve_dev(NULL, d, d, 1, 1, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_solve_banded_dev(NULL, d, d, 1, 1, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_sample_dev(NULL, d, d, i32, i64, 1, 1, 0.01, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_sample_yaw_dev(NULL, d, d, i32, i64, 1, 1, 0.01, d, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_sample_hits_dev(NULL, d, d, i32, i64, 1, 1, 0.01, d, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_sample_derivs_dev(NULL, d, i32, i64, 1, 1, 0.01, d, d, d, d, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_plan_dev(NULL, d, 1, 1, 1.0, 0.01, d, i32, i64, d, i32, d, 1, d, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_row_offsets_dev(NULL, i32, 1, 1, i64) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_row_offsets_ragged_dev(NULL, i32, i64, 1, 1, i64) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_plan_ragged(NULL, d, i64, 1, 1.0, 0.01, d, i64, d, d, 0) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_row_counts_ragged_dev(NULL, d, i64, 1, 1, 1.0, 0.01, d, i32, i64) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_solve_ragged_dev(NULL, d, d, i64, 1, 1, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_sample_ragged_dev(NULL, d, i32, i64, i64, 1, 1, 1, 0.01, d, 1, d, i32, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_obstacle_round_dev(NULL, d, i64, 1, 1, 1.0, 0.01, d, i32, i32, i32, d, i64, i32, d, i32, i64, d, i32) ==
           UAVAC_EINVAL);
    EXPECT(uavac_minsnap_obstacle_waypoints(NULL, d, i64, 1, 1.0, 0.01, d, 1, 4, 0, d, 8, i64, i32) == UAVAC_EINVAL);
    EXPECT(uavac_yaw_scan_dev(NULL, d, i64, 1, d) == UAVAC_EINVAL && uavac_yaw_scan(NULL, d, 1, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_row_counts(NULL, d, 1, 1, 1.0, 0.01, d, i32, i64) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_solve(NULL, d, 1, 1, 1.0, d, d) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_sample(NULL, d, d, 1, 1, 0.01, i64, d) == UAVAC_EINVAL);
    EXPECT(uavac_state_init_dev(NULL, &V, d, 1, 1, d, i32) == UAVAC_EINVAL && uavac_state_init(NULL, &V, d, 1, 1, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_control_rollout_dev(NULL, &V, d, i64, d, i32, 1, 1, d, d, d, 0) == UAVAC_EINVAL);
    EXPECT(uavac_control_rollout_plan_dev(NULL, &V, d, i32, i64, d, d, 1, 0.01, d, i32, 1, 1, d, d, d, 0) == UAVAC_EINVAL);
    EXPECT(uavac_control_rollout_plan_ragged_dev(NULL, &V, d, i32, i64, i64, d, 1, 0.01, d, i32, 1, 1, d, d, d, 0) == UAVAC_EINVAL);
    EXPECT(uavac_control_step_dev(NULL, &V, d, i64, d, i32, 1) == UAVAC_EINVAL);
    EXPECT(uavac_control_rollout(NULL, &V, d, i64, d, i32, 1, 1, d, d, d, 0) == UAVAC_EINVAL);
    EXPECT(uavac_controller_tick_dev(NULL, &V, d, i64, d, i32, 1) == UAVAC_EINVAL);
    EXPECT(uavac_dynamics_step_dev(NULL, &V, d, i32, 1, d, 0) == UAVAC_EINVAL);
    EXPECT(uavac_controller_tick(NULL, &V, d, i64, d, i32, 1) == UAVAC_EINVAL);
    EXPECT(uavac_dynamics_step(NULL, &V, d, i32, 1, d, 0) == UAVAC_EINVAL);
    uavac_pilot *pilot = (uavac_pilot *)0x1;
    EXPECT(uavac_pilot_create(NULL, d, i64, 1, &pilot) != UAVAC_OK);
    uavac_pilot_destroy(NULL);
    EXPECT(uavac_pilot_state(NULL) == NULL && uavac_pilot_istate(NULL) == NULL);
    EXPECT(uavac_pilot_set_obstacles(NULL, d, 1) != UAVAC_OK && uavac_pilot_tick(NULL, &V, 3) != UAVAC_OK);
    EXPECT(uavac_probe_outer(NULL, &V, d, 1, 0, d) == UAVAC_EINVAL && uavac_probe_inner(NULL, &V, d, 1, 0, d) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_star_dev(NULL, d, d, 1, 1.0, 1, d, d, 0, d, i32, i32, i32, d, i32, d) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_star(NULL, d, d, 1, 1.0, 1, d, d, 0, d, i32, i32, i32, d, i32, d) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_segment_hits_dev(NULL, d, d, 1, d, 1, i32) == UAVAC_EINVAL && uavac_rrt_segment_hits(NULL, d, d, 1, d, 1, i32) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_edge_lengths_dev(NULL, d, d, 0, 1, d) == UAVAC_EINVAL && uavac_rrt_edge_lengths(NULL, d, d, 0, 1, d) == UAVAC_EINVAL);
    uint32_t seeds[1] = {1};
    EXPECT(uavac_rrt_draw_nodes_dev(NULL, seeds, d, 1, 1, d, d, 0.15, d, i64) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_simplify_dev(NULL, d, i32, 1, 2, d, 0, d, i32) == UAVAC_EINVAL && uavac_rrt_simplify(NULL, d, i32, 1, 2, d, 0, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_path_cost_dev(NULL, d, 2, d) == UAVAC_EINVAL && uavac_rrt_path_cost(NULL, d, 2, d) == UAVAC_EINVAL);
    EXPECT(uavac_rrt_steer_dev(NULL, d, d, 1, 1.0, d) == UAVAC_EINVAL && uavac_rrt_steer(NULL, d, d, 1, 1.0, d) == UAVAC_EINVAL);
    char id[UAVAC_COMM_ID_BYTES];
    void *comm = NULL;
    int world = 0, rank = 0;
    EXPECT(uavac_comm_unique_id(NULL, id) == UAVAC_EINVAL && uavac_comm_init_rank(NULL, id, 1, 0, &comm) == UAVAC_EINVAL);
    EXPECT(uavac_comm_destroy(NULL, NULL) == UAVAC_EINVAL && uavac_comm_abort(NULL, NULL) == UAVAC_EINVAL);
    EXPECT(uavac_comm_shape(NULL, NULL, &world, &rank) == UAVAC_EINVAL && uavac_gather_counts(NULL, NULL, 1, i64) == UAVAC_EINVAL);
    EXPECT(uavac_gather_rows_dev(NULL, NULL, d, 1, 11, i64, 0, d) == UAVAC_EINVAL);
    EXPECT(uavac_gather_plan_dev(NULL, NULL, d, d, i32, 1, i64, 0, d, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_gather_plan_part_dev(NULL, NULL, d, d, i32, i64, i64, i64, 0, d, d, i32) == UAVAC_EINVAL);
    EXPECT(uavac_minsnap_first_yaw_dev(NULL, d, i32, NULL, 1, 1, 0.01, d) == UAVAC_EINVAL);
    EXPECT(uavac_comm_finish(NULL, NULL) == UAVAC_EINVAL && uavac_comm_loopback_dev(NULL, NULL, d, d, 1) == UAVAC_EINVAL);
    printf("asan driver: %d entry points answered a GPU-less host as documented\n", 74);
    return 0;
}
