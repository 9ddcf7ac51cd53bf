// Monte-Carlo k-NN kernels (knn.py) -- see fz_knn_host.inc for the ABI side.
#pragma once
#include "fz_device.h"
