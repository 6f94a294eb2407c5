// empty stand-in for <opencv2/features2d.hpp> (tests/stubs/README.md): the interface headers include it but name nothing of it
#pragma once
