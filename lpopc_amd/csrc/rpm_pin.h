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
  RPM_PIN_OVERLAP_REFUSED,       /* requests refused because they partly overlap pages another engine holds */
  RPM_PIN_SHARED,                /* requests served by a registration another holder made */
  RPM_PIN_MERGED,                /* requests that replaced the requester's own overlapping registrations */
  RPM_PIN_EVICTED,               /* holds dropped because an engine exceeded its cap */
  RPM_PIN_N_COUNTERS,
  RPM_PIN_LIVE = 100             /* registrations in the table right now */
};

/* Device-visible alias of `ptr` (valid for [ptr, ptr + bytes)) after making sure its pages are registered and `owner`
 * holds them, or NULL: below the threshold, refused by the runtime, or overlapping another holder's pages.  An owner keeps
 * at most max_holds registrations (least recently used goes first). */
void* rpm_pin_acquire(const void* owner, const void* ptr, size_t bytes, int max_holds);
void rpm_pin_release_owner(const void* owner);                                 /* every hold of `owner` */
int rpm_pin_release_range(const void* owner, const void* ptr, size_t bytes);   /* its holds that touch the range; count */
long rpm_pin_counter(int which);
int rpm_pin_held(const void* owner);
size_t rpm_pin_last_error(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
