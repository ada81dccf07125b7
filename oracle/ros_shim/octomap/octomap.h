#pragma once
// TEST INFRASTRUCTURE: empty stand-in (the reference includes this header and uses nothing of it on the compiled path)
