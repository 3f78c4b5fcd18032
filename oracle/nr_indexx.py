"""TEST INFRASTRUCTURE (oracle): the reference's index sort `indexx_i4b`, route/build/src/nr_utils.f90:114-177, restated
line for line -- quicksort on the index with median of three, insertion sort below 15 elements, NOT stable.  `assign_node`
(domain_decomposition.f90:724-819) ranks the domains with it, so which of two equally large domains goes to which node
follows its order; the partition tests pass this routine to `partition.reference_domains(sort_index=...)` to compare node
for node with the compiled reference (oracle/_ref/ref_topo).  The product does not use it (a stable argsort there)."""
import numpy as np


def nr_indexx(arr) -> np.ndarray:
    """Index array that sorts `arr` ascending, element for element what the reference's `indexx` returns
    (nr_utils.f90:114-190: quicksort on the index with median of three, insertion sort below 15 elements; NOT stable, and
    assign_node's treatment of equally large domains follows its order).  0-based result."""
    a = np.asarray(arr)
    n = a.size
    idx = list(range(n + 1))                 # 1-based like the source: idx[1..n] hold 1-based positions
    A = lambda i: a[i - 1]
    NN = 15
    stack = []
    l, r = 1, n
    while True:
        if r - l < NN:
            for j in range(l + 1, r + 1):
                indext = idx[j]
                av = A(indext)
                i = j - 1
                while i >= 1:
                    if A(idx[i]) <= av:
                        break
                    idx[i + 1] = idx[i]
                    i -= 1
                idx[i + 1] = indext
            if not stack:
                break
            r = stack.pop(); l = stack.pop()
        else:
            k = (l + r) // 2
            if k != l + 1:
                idx[k], idx[l + 1] = idx[l + 1], idx[k]
            if A(idx[r]) < A(idx[l]):
                idx[l], idx[r] = idx[r], idx[l]
            if A(idx[r]) < A(idx[l + 1]):
                idx[l + 1], idx[r] = idx[r], idx[l + 1]
            if A(idx[l + 1]) < A(idx[l]):
                idx[l], idx[l + 1] = idx[l + 1], idx[l]
            i, j = l + 1, r
            indext = idx[l + 1]
            av = A(indext)
            while True:
                i += 1
                while A(idx[i]) < av:
                    i += 1
                j -= 1
                while A(idx[j]) > av:
                    j -= 1
                if j < i:
                    break
                if i != j:
                    idx[i], idx[j] = idx[j], idx[i]
            idx[l + 1] = idx[j]
            idx[j] = indext
            if r - i + 1 >= j - l:
                stack.append(i); stack.append(r)
                r = j - 1
            else:
                stack.append(l); stack.append(j - 1)
                l = i
    return np.array(idx[1:], dtype=np.int64) - 1
