// Host-side data-format helpers of libuoc_hip.so (no device code): the LZF decoder the dataset loaders need for
// `DATA binary_compressed` point clouds.  The reference reads OCID / OSD clouds through python-pcl
// (/root/reference/lib/datasets/ocid_object.py:105, osd_object.py:92); PCL stores such files as one LZF stream
// (liblzf format) of the fields in structure-of-arrays order.
#include "common.h"

extern "C" {

// liblzf stream format: ctrl < 32 -> a literal run of ctrl+1 bytes; otherwise a back reference of length
// (ctrl >> 5) + 2 (length code 7 takes one more length byte) at distance ((ctrl & 31) << 8 | next byte) + 1.
long uoc_lzf_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap) {
  if (!in || !out) {
    uoc::set_error("lzf: null pointer");
    return UOC_EINVAL;
  }
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_cap) {
        uoc::set_error("lzf: literal run overflows (in %zu/%zu, out %zu/%zu)", ip + run, in_len, op + run, out_cap);
        return UOC_EINVAL;
      }
      for (size_t i = 0; i < run; ++i) out[op++] = in[ip++];
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) {
        if (ip >= in_len) {
          uoc::set_error("lzf: truncated stream");
          return UOC_EINVAL;
        }
        len += in[ip++];
      }
      if (ip >= in_len) {
        uoc::set_error("lzf: truncated stream");
        return UOC_EINVAL;
      }
      const size_t dist = (((size_t)(ctrl & 0x1f)) << 8) + in[ip++] + 1;
      len += 2;
      if (dist > op || op + len > out_cap) {
        uoc::set_error("lzf: bad back reference (distance %zu at %zu, length %zu, capacity %zu)", dist, op, len, out_cap);
        return UOC_EINVAL;
      }
      for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];  // may overlap: byte by byte
    }
  }
  return (long)op;
}

}  // extern "C"
