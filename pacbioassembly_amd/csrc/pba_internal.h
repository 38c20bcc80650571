// pba_internal.h -- shared between the host translation units of libpba.so.
#ifndef PBA_INTERNAL_H
#define PBA_INTERNAL_H

#include <stdint.h>

// C2I, /root/reference/src/dna_seq.h:21: A,C,G -> 0,1,2; any other byte -> 3
static inline unsigned pba_c2i(unsigned ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u; }

#endif
