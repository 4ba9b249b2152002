/* ssd_phase_readers.h — tools builds only (-DSSD_PHASE_TIMING): how tools/phases.py and tools/blockphases.py read the clocks of
 * ssd_phase.h.  Included at the end of ssd_kernels.hip, outside namespace ssd; empty in the product build. */
#ifdef SSD_PHASE_TIMING
extern "C" __attribute__((visibility("default"))) int ssd_phase_read(unsigned long long *out)
{
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(ssd::g_phase), sizeof(ssd::g_phase)));
}
/* reads and clears the block-life accumulators of the streaming kernels (tools/blockphases.py) */
extern "C" __attribute__((visibility("default"))) int ssd_blockphase_read(unsigned long long *out)
{
  const int rc = static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(ssd::g_blockphase), sizeof(ssd::g_blockphase)));
  static const unsigned long long zero[4][64][8] = {};
  return rc ? rc : static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(ssd::g_blockphase), zero, sizeof(zero)));
}
#endif
