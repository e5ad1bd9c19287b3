/* rpm_pin.h — interface of librpm_pin.so, the process-wide table of page-locked caller memory (rpm_pin.cpp).  Internal:
 * librpm_hip.so and the user-problem libraries call it; callers of include/rpm_hip.h see it through the option "pin_host"
 * and the get-only options "pin_*". */
#ifndef RPM_PIN_H_
#define RPM_PIN_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RPM_PIN_MIN_BYTES (64u * 1024u)   /* smaller arrays are never registered */

enum {
  RPM_PIN_REGISTERED = 0,        /* hipHostRegister calls that succeeded */
  RPM_PIN_REGISTER_FAILURES,     /* ... that were refused by the runtime */
  RPM_PIN_UNREGISTERED,          /* hipHostUnregister calls that succeeded */
  RPM_PIN_UNREGISTER_FAILURES,   /* ... that were refused */
  RPM_PIN_OVERLAP_REFUSED,       /* requests refused because they partly overlap memory another engine holds */
  RPM_PIN_SHARED,                /* requests that joined a region another engine holds arrays in */
  RPM_PIN_MERGED,                /* requests that overlapped the requester's own regions (one region now covers them) */
  RPM_PIN_EVICTED,               /* arrays an engine let go of because it reached its cap */
  RPM_PIN_N_COUNTERS,
  RPM_PIN_LIVE = 100,            /* regions registered right now */
  RPM_PIN_LIVE_BYTES = 101       /* ... and their bytes */
};

/* Device-visible alias of `ptr` (valid for [ptr, ptr + bytes)) after making sure its pages are registered and `owner`
 * holds the array, or NULL: below the threshold, refused by the runtime, or partly overlapping memory another owner
 * holds.  An owner keeps at most max_holds arrays (least recently used goes first).  Call it for a new array only while
 * nothing of the owner is in flight: the owner's regions may be re-registered.  also_small: register arrays below
 * RPM_PIN_MIN_BYTES too (an interval-sharded engine stores only ITS rows into the caller's array: no staged copy can stand in). */
void* rpm_pin_acquire(const void* owner, const void* ptr, size_t bytes, int max_holds, int also_small);
void rpm_pin_release_owner(const void* owner);                                 /* every array of `owner` */
long rpm_pin_counter(int which);
int rpm_pin_held(const void* owner);
size_t rpm_pin_last_error(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
