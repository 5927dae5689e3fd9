"""Print selected functions of a cProfile dump: python tools/prof_callers.py FILE pattern [pattern ...]"""
import pstats
import sys
st = pstats.Stats(sys.argv[1])
st.sort_stats("cumulative")
for pat in sys.argv[2:]:
    st.print_stats(pat)
