/*
 * antiqsort.cpp — M. D. McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) run against this platform's
 * std::sort: produces a permutation of 0..n-1 on which std::sort's quicksort phase degenerates, so that its depth limit
 * (2 * lg n) is exhausted and the heapsort fallback runs.  TEST INFRASTRUCTURE: tests use the sequence (optionally
 * quantised to create ties) to drive csrc/ssd_sort.h through that path and compare it with std::sort itself.
 * usage: antiqsort n  ->  prints n integers
 */
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

static std::vector<int> val;     /* value of item i, or gas */
static int nsolid, candidate, gas;

static bool cmp(int x, int y)
{
  if(val[x] == gas && val[y] == gas)
  {
    if(x == candidate) val[x] = nsolid++;
    else val[y] = nsolid++;
  }
  if(val[x] == gas) candidate = x;
  else if(val[y] == gas) candidate = y;
  return val[x] < val[y];
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 300;
  val.assign(size_t(n), 0);
  std::vector<int> ptr(static_cast<size_t>(n));
  gas = n - 1;
  nsolid = candidate = 0;
  for(int i = 0; i < n; i++) { ptr[size_t(i)] = i; val[size_t(i)] = gas; }
  std::sort(ptr.begin(), ptr.end(), cmp);
  for(int i = 0; i < n; i++) printf("%d\n", val[size_t(i)]);
  return 0;
}
