/*
 * ssd_closing.h — cv::morphologyEx(MORPH_CLOSE, 3x3 rect, 1 iteration, default border; segmentation.cpp:888,928) on the bit
 * images of the kernels: whole 64-bit words on the fly (closed_word / closed_column: the probe rows of k_outline, the debug
 * capture) and single pixel columns (closed_scan_column: the scans of k_outline and k_final).  Host + device code: the
 * kernels run the device build, the CPU test suite the host build (ssd_test_closing_host, ssd_testhooks.hip) against the
 * oracle's closing, which is itself checked against scipy.ndimage.
 */
#ifndef SSD_CLOSING_H_
#define SSD_CLOSING_H_

#include <hip/hip_runtime.h>
#include <cstddef>

namespace ssd
{

struct BitImg
{
  const unsigned long long *w;   /* H rows of W64 words; bit i of word c is pixel x = 64c + i */
  int W, H, W64;
  const unsigned long long *w2 = nullptr;   /* a second image of the same shape, OR-ed in (single pass: the planes of a plateau's two bins) */
};

__host__ __device__ __forceinline__ unsigned long long raw_word(const BitImg &im, int y, int c)
{
  if(!(y >= 0 && y < im.H && c >= 0 && c < im.W64))
    return 0ull;
  const size_t at = static_cast<size_t>(y) * im.W64 + c;
  unsigned long long v = im.w[at];
  if(im.w2)
    v |= im.w2[at];
  return v;
}

/* word c of row y of the image after cv::morphologyEx(MORPH_CLOSE, 3x3 rect, 1 iteration, default
 * border): dilation = OR over the in-image 3x3 neighbours, erosion = AND over them (pixels outside
 * the image never win; segmentation.cpp:888,928).  Needs raw rows y-2..y+2, words c-1..c+1. */
struct HRow
{
  unsigned long long c;          /* word c of a row, dilated horizontally */
  unsigned int l, r;             /* the dilated pixels x = 64c-1 and x = 64c+64 */
};

__host__ __device__ __forceinline__ HRow hdilated_row(const BitImg &im, int yy, int c)
{
  const unsigned long long L = raw_word(im, yy, c - 1), w = raw_word(im, yy, c), R = raw_word(im, yy, c + 1);
  HRow h;
  h.c = w | (w << 1) | (w >> 1) | (L >> 63) | (R << 63);
  h.l = static_cast<unsigned int>(((L >> 63) | (L >> 62) | w) & 1ull);
  h.r = static_cast<unsigned int>((R | (R >> 1) | (w >> 63)) & 1ull);
  return h;
}

/* h[r] = hdilated_row(y - 2 + r) */
__host__ __device__ __forceinline__ unsigned long long closed_from_rows(const BitImg &im, int y, int c, const HRow h[5])
{
  const int rem = im.W - 64 * c;
  const unsigned long long vm = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);        /* pixels of this word inside the image */
  unsigned long long res = vm;
#pragma unroll
  for(int r = 0; r < 3; r++)
  {
    const int yy = y - 1 + r;
    if(yy < 0 || yy >= im.H)
      continue;                                   /* row outside the image: ignored by the erosion */
    unsigned long long dc = h[r].c | h[r + 1].c | h[r + 2].c;
    unsigned int dl = h[r].l | h[r + 1].l | h[r + 2].l;
    unsigned int dr = h[r].r | h[r + 1].r | h[r + 2].r;
    dc |= ~vm;
    if(c == 0) dl = 1u;
    if(rem <= 64) dr = 1u;
    res &= dc & ((dc << 1) | dl) & ((dc >> 1) | (static_cast<unsigned long long>(dr) << 63));
  }
  return res;
}

/* a bounding box of raw bits (rows by0..by1, word columns ..bc1; word column 0 holds x = 0 and x = 1 alike) ->
 * the box its closing can reach: one more row / the last word column when the bits come within one pixel of
 * the image border, where the erosion has no out-of-image neighbour to veto it */
__host__ __device__ __forceinline__ void grow_box_to_border(int W, int H, int W64, int &by0, int &by1, int &bc1)
{
  if(by1 < by0)
    return;
  if(by0 == 1)
    by0 = 0;
  if(by1 == H - 2)
    by1 = H - 1;
  if(bc1 == W64 - 2 && ((W - 1) & 63) == 0)
    bc1 = W64 - 1;
}

__host__ __device__ inline unsigned long long closed_word(const BitImg &im, int y, int c)
{
  HRow h[5];
#pragma unroll
  for(int r = 0; r < 5; r++)
    h[r] = hdilated_row(im, y - 2 + r, c);
  return closed_from_rows(im, y, c, h);
}

/* closed words of column c, rows [yA, yB), top to bottom with a rolling window of five dilated rows
 * (3 word loads per row instead of 15); calls visit(y, closedWord) for the non-zero ones (all with `all`) */
template<typename Visit>
__host__ __device__ __forceinline__ void closed_column(const BitImg &im, int c, int yA, int yB, bool all, Visit visit)
{
  HRow h[5];
#pragma unroll
  for(int r = 1; r < 5; r++)
    h[r] = hdilated_row(im, yA - 3 + r, c);
  /* four rows per step: their twelve word loads are issued together (one memory round trip instead of four — the
   * walk is a chain of dependent latencies, and with outliers the box is the whole image) */
  for(int y0 = yA; y0 < yB; y0 += 4)
  {
    HRow next[4];
#pragma unroll
    for(int k = 0; k < 4; k++)
      next[k] = hdilated_row(im, y0 + k + 2, c);          /* rows beyond the image read as zero */
#pragma unroll
    for(int k = 0; k < 4; k++)
    {
      const int y = y0 + k;
#pragma unroll
      for(int r = 0; r < 4; r++)
        h[r] = h[r + 1];
      h[4] = next[k];
      if(y >= yB)
        continue;
      if(!all && (h[0].c | h[1].c | h[2].c | h[3].c | h[4].c) == 0ull)
        continue;                                 /* nothing lit within two rows: the closing has nothing either */
      visit(y, closed_from_rows(im, y, c, h));
    }
  }
}

/* The closed image at ONE pixel column x, rows [yA, yB) top to bottom: hit(y) for every closed pixel (x, y).
 * The scans of the outline want nothing else of the closed image than its first and last lit row in every 25th (50th)
 * column, and the closing at (x, y) is a function of the raw 5 x 5 neighbourhood only: per row a 5-bit strip of the raw
 * row (pixels x-2 .. x+2; two 32-bit loads and a funnel shift), its horizontal dilation at x-1, x, x+1 (3 bits), then the
 * vertical dilation and the erosion on those 3-bit values — a dozen 32-bit operations per row, against some hundred 64-bit
 * ones for a whole closed word.  Same definition as closed_from_rows: pixels outside the image never veto the erosion and
 * never feed the dilation (cv::morphologyEx, default border; segmentation.cpp:888,928). */
/* bits sh .. sh + 31 of hi:lo (v_alignbit_b32 on the device) */
__host__ __device__ __forceinline__ unsigned int funnel_right(unsigned int lo, unsigned int hi, int sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __funnelshift_r(lo, hi, sh);
#else
  return sh == 0 ? lo : (lo >> sh) | (hi << (32 - sh));
#endif
}

/* A scan column's view of the image: per row the five pixels x-2 .. x+2.  They lie in the 32-bit word of pixel x-2 and at most
 * the next one.  Two ways to fetch them (strip_hdil<DUAL, PAIR>):
 *  - two 4-byte loads and a funnel shift: the fewest vector instructions;
 *  - PAIR: ONE 8-byte load (4-byte aligned) of the word pair at `pair`, the five pixels by a 64-bit shift - half the load
 *    instructions of a band.  The pair never leaves the row: at the row's last word it starts one word earlier (32 more bits
 *    shifted out), left of the image (x < 2) it starts at word 0 and the value is shifted LEFT (zeros for the pixels outside).
 * k_outline waits for the loads of its bands when the images are large and outliers stretch the boxes over them (FHD stress: 0.341 ->
 * 0.295 ms per 256 frames with PAIR) and for its arithmetic when they are not (XGA: 0.140 -> 0.147 ms): PAIR for images wider than
 * 1024 pixels (closed_scan_column). */
struct __attribute__((packed, aligned(4))) WordPair { unsigned int lo, hi; };
struct ColumnStrip
{
  const unsigned int *row0;      /* the image as 32-bit words */
  const unsigned int *row0b;     /* BitImg::w2 likewise, or null */
  long long stride;              /* 32-bit words per row */
  int i0, i1, sh, H;             /* word indices of pixel x-2 and of the word after it (-1: outside), shift of pixel x-2 */
  int pair, left, right;         /* PAIR: first word of the pair; shifts that bring pixel x-2 to bit 0 */
  unsigned int ignore;           /* of x-1, x, x+1 the positions outside the image */
};
__host__ __device__ __forceinline__ ColumnStrip column_strip(const BitImg &im, int x)
{
  ColumnStrip c;
  c.row0 = reinterpret_cast<const unsigned int *>(im.w);
  c.row0b = reinterpret_cast<const unsigned int *>(im.w2);
  c.stride = 2ll * im.W64;
  const int xs = x - 2;
  const int w = xs >> 5;                               /* arithmetic: -1 for xs < 0 */
  c.sh = xs & 31;
  c.i0 = (w >= 0 && w < 2 * im.W64) ? w : -1;
  c.i1 = (w + 1 >= 0 && w + 1 < 2 * im.W64) ? w + 1 : -1;
  const int last = 2 * im.W64 - 2;                     /* the last word a pair can start at (W64 >= 1) */
  c.pair = w < 0 ? 0 : (w > last ? last : w);
  const int shift = c.sh + 32 * (w - c.pair);          /* -2, -1 (x = 0, 1) or 0 .. 63 */
  c.left = shift < 0 ? -shift : 0;
  c.right = shift > 0 ? shift : 0;
  c.H = im.H;
  c.ignore = (x - 1 < 0 ? 1u : 0u) | (x + 1 >= im.W ? 4u : 0u);
  return c;
}
/* horizontally dilated raw row yy at x-1, x, x+1 (3 bits); rows outside the image read as zero */
template<bool DUAL, bool PAIR>
__host__ __device__ __forceinline__ unsigned int strip_hdil(const ColumnStrip &c, int yy, int yEnd)
{
  const bool in = yy >= 0 && yy < c.H && yy < yEnd;
  unsigned int v;
  if(PAIR)
  {
    const long long at = static_cast<long long>(in ? yy : 0) * c.stride + c.pair;
    WordPair p = *reinterpret_cast<const WordPair *>(c.row0 + at);
    if(DUAL)
    {
      const WordPair q = *reinterpret_cast<const WordPair *>(c.row0b + at);
      p.lo |= q.lo;
      p.hi |= q.hi;
    }
    const unsigned long long both = in ? (static_cast<unsigned long long>(p.hi) << 32) | p.lo : 0ull;
    v = (static_cast<unsigned int>(both >> c.right) << c.left) & 31u;      /* left > 0 only with right == 0 */
  }
  else
  {
    const unsigned int *r = c.row0 + static_cast<long long>(in ? yy : 0) * c.stride;
    unsigned int lo = (in && c.i0 >= 0) ? r[c.i0] : 0u;
    unsigned int hi = (in && c.i1 >= 0) ? r[c.i1] : 0u;
    if(DUAL)
    {
      const unsigned int *rb = c.row0b + static_cast<long long>(in ? yy : 0) * c.stride;
      lo |= (in && c.i0 >= 0) ? rb[c.i0] : 0u;
      hi |= (in && c.i1 >= 0) ? rb[c.i1] : 0u;
    }
    v = funnel_right(lo, hi, c.sh) & 31u;
  }
  return (v | (v >> 1) | (v >> 2)) & 7u;
}
/* DUAL: two images read together - decided once per call, so that the loads of a step's rows are still issued together
 * (a test of the second pointer inside strip_hdil put a branch between every two of them: K3 0.12 -> 0.21 ms) */
template<bool DUAL, bool PAIR, typename Hit>
__host__ __device__ __forceinline__ void closed_scan_column_impl(const BitImg &im, int x, int yA, int yB, Hit hit)
{
  constexpr int kRows = 16;                            /* rows per step: their loads are issued together */
  const int yEnd = yB + 2;                             /* rows from here on feed no row of the band: not loaded */
  const ColumnStrip c = column_strip(im, x);
  /* h[i] = the dilated strip of row y0 - 2 + i.  The four rows above the band are loaded with the first step's rows:
   * nothing is consumed before all of them are under way (a band is a chain of memory round trips and little else) */
  unsigned int h[kRows + 4];
#pragma unroll
  for(int k = 0; k < 4; k++)
    h[k] = strip_hdil<DUAL, PAIR>(c, yA - 2 + k, yEnd);
  for(int y0 = yA; y0 < yB; y0 += kRows)
  {
#pragma unroll
    for(int k = 0; k < kRows; k++)
      h[4 + k] = strip_hdil<DUAL, PAIR>(c, y0 + 2 + k, yEnd);
    /* nothing lit within two rows of the step's rows (the common case where a few outlier pixels have stretched the bounding
     * box over an otherwise empty image): no closed pixel either — a closed pixel needs its own row's dilation lit */
    unsigned int any = 0u;
#pragma unroll
    for(int k = 0; k < kRows + 4; k++)
      any |= h[k];
    if(any == 0u)
      continue;                                          /* (the rows carried over to the next step are zero as well) */
    /* f bit i: the dilated image is lit at every in-image pixel of x-1 .. x+1 in row y0 - 2 + i (rows outside: set) */
    unsigned int f = 0u;
#pragma unroll
    for(int i = 1; i < kRows + 3; i++)
    {
      const int yy = y0 - 2 + i;
      const bool full = yy < 0 || yy >= c.H || ((h[i - 1] | h[i] | h[i + 1] | c.ignore) & 7u) == 7u;
      f |= full ? 1u << i : 0u;
    }
#pragma unroll
    for(int k = 0; k < kRows; k++)
      if(y0 + k < yB && ((f >> (k + 1)) & 7u) == 7u)
        hit(y0 + k);
#pragma unroll
    for(int k = 0; k < 4; k++)
      h[k] = h[kRows + k];
  }
}
template<typename Hit>
__host__ __device__ __forceinline__ void closed_scan_column(const BitImg &im, int x, int yA, int yB, Hit hit)
{
  const bool pair = im.W64 > 16;                       /* wider than 1024 pixels: see ColumnStrip */
  if(im.w2)
  {
    if(pair)
      closed_scan_column_impl<true, true>(im, x, yA, yB, hit);
    else
      closed_scan_column_impl<true, false>(im, x, yA, yB, hit);
  }
  else
  {
    if(pair)
      closed_scan_column_impl<false, true>(im, x, yA, yB, hit);
    else
      closed_scan_column_impl<false, false>(im, x, yA, yB, hit);
  }
}

} // namespace ssd

#endif /* SSD_CLOSING_H_ */
