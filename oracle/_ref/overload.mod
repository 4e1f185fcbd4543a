﻿!mod$ v1 sum:50fc1fe729f71194
module overload
end
