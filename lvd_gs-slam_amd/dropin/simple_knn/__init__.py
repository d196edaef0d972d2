"""`simple_knn` (reference README.md:42) served by lvdgs."""
