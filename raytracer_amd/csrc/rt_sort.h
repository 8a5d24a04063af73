// rt_sort.hip: stable radix sort of (key, value) pairs on the device.  tmp == NULL: only *tmp_bytes is set (the temporary
// storage the sort needs for n pairs); otherwise the sort is enqueued on `stream`.  Keys are compared on bits [0, end_bit).
#pragma once
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
int rt_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                      const unsigned int* vals_in, unsigned int* vals_out, size_t n, int end_bit, void* stream);
#ifdef __cplusplus
}
#endif
