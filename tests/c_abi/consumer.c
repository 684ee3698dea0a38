/* TEST INFRASTRUCTURE: a plain-C consumer of include/dynaboa_hip.h - what a maintainer binding the hot path from C / cgo / JNI
 * writes first ("The per-frame stepper from another host language").  No compute: it creates the engine plan and
 * a frame stepper, sets options by name, reads the sizes a caller must allocate, and tears both down.  tests/test_host_emu.py
 * compiles it with gcc, links it against the library under test and compares what it prints with the Python binding. */
#include <stdio.h>
#include "dynaboa_hip.h"

int main(void) {
  void *plan = 0, *st = 0;
  if (dyb_hmr_plan_create(1, 224, 224, &plan) != 0 || !plan) return 2;
  printf("param_floats %zu\n", dyb_hmr_param_floats(plan));
  printf("act_floats %zu\n", dyb_hmr_act_floats(plan));
  printf("workspace_bytes %zu\n", dyb_hmr_workspace_bytes(plan));
  printf("tensors %d\n", dyb_hmr_num_tensors(plan));
  printf("hvp_dual_floats %zu\n", dyb_hmr_hvp_dual_floats(plan));
  if (dyb_stepper_create(plan, 1, 224, 224, &st) != 0 || !st) return 3;
  if (dyb_stepper_set_i(st, "inner_step", 3) != 0) return 4;
  if (dyb_stepper_set_f(st, "lr", 3e-6) != 0) return 5;
  if (dyb_stepper_set_i(st, "replicas", 4) != 0) return 6;
  if (dyb_stepper_set_i(st, "no_such_option", 1) == 0) return 7;          /* unknown keys are errors, not silently ignored */
  printf("record_floats %lld\n", dyb_stepper_get_i(st, "record_floats"));
  printf("loss_floats %lld\n", dyb_stepper_get_i(st, "loss_floats"));
  printf("stepper_workspace_bytes %zu\n", dyb_stepper_workspace_bytes(st));
  dyb_stepper_destroy(st);
  dyb_hmr_plan_destroy(plan);
  return 0;
}
