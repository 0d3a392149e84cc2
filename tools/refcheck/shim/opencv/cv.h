#include "refcheck_cv.h"
