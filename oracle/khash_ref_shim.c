/* TEST INFRASTRUCTURE ONLY (oracle/).  Builds the reference's own native component
 * (klib khash int64->int64 map, /root/reference/lib/khash.h + khash_int2int.h) from the
 * sources where they lie, into oracle/_ref/libkhash_ref.so.  Nothing is copied: this file only
 * includes the reference header and re-exports its four `static` entry points
 * (khash_int2int.h:8-33) under non-static names so ctypes can bind them.
 * Used (a) by tests/golden/make_golden.py to run the reference python in this container
 * (stands in for the cffi module `_khash_ffi`, whose runtime `_cffi_backend` is absent),
 * (b) by tests to cross-check oracle/lattice_oracle.c's own hash. */
#include "khash_int2int.h"

void *ref_khash_int2int_init(void) { return khash_int2int_init(); }
void ref_khash_int2int_destroy(void *h) { khash_int2int_destroy(h); }
long long ref_khash_int2int_get(void *h, long long key, long long dflt) {
    return (long long)khash_int2int_get(h, (khint64_t)key, (khint64_t)dflt);
}
int ref_khash_int2int_set(void *h, long long key, long long val) {
    return khash_int2int_set(h, (khint64_t)key, (khint64_t)val);
}
