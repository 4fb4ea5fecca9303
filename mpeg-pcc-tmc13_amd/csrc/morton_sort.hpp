// morton_sort.hpp -- Morton encode + stable radix sort (placeholder; the
// kernels land in a later commit of this round).
#pragma once
