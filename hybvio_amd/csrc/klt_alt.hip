// A/B build of klt.hip for measurements in one process (knob klt_tile 6 / 7): the same source compiled with KLT_ALT, which switches on
// the edits under test (the #ifdef KLT_ALT blocks of klt.hip).
#define KLT_ALT 1
#define launch_klt launch_klt_alt
#include "klt.hip"
