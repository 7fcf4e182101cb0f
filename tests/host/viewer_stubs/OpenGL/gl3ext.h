// test aid: see gl3.h
#pragma once
