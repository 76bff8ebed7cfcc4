// TEST INFRASTRUCTURE: sample_app/main.cpp includes this header and uses nothing from it.
#pragma once
