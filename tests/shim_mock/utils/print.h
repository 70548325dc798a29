// stand-in for ov_core/src/utils/print.h (TEST INFRASTRUCTURE)
#pragma once
#include <cstdio>
#define PRINT_DEBUG(...) ((void)0)
#define PRINT_INFO(...) ((void)0)
#define PRINT_WARNING(...) ((void)0)
#define PRINT_ERROR(...) ((void)0)
#define RED ""
#define RESET ""
