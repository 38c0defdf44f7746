// r3o_tally.cpp -- TEST INFRASTRUCTURE: the oracle's restatement (r3o.c, untouched) compiled with a counting f32 (tally.h).
// Built by oracle/lib.py::build_tally into oracle/libr3o_tally.so (git-ignored); loaded only by bench.py's roofline leg and
// tests/test_oracle_goldens.py.
#include "tally.h"
extern "C" {
unsigned long long r3o_tally_all[3][8];
unsigned long long *r3o_tally = r3o_tally_all[0];
void r3o_tally_reset(void) { memset(r3o_tally_all, 0, sizeof r3o_tally_all); r3o_tally = r3o_tally_all[0]; }
void r3o_tally_read(unsigned long long out[24]) { memcpy(out, r3o_tally_all, sizeof r3o_tally_all); }
#include "r3o.c"
}
